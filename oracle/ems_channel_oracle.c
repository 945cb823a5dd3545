/* ems_channel_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * Plain-C restatement of the reference's constellation channel front ends, every float /
 * double promotion spelt out as gcc evaluates the reference's expressions on x86-64 (FLT_EVAL_METHOD 0):
 *   ModelChannel_AWGN_64          channel.c:112-310   64-APSK (DVB-S2X 8+16+20+20), AWGN, GF(64)
 *   ModelChannel                  channel.c:328-736   as the shipped switches compile it (channel.c:20-26): rotated QAM
 *                                                     (QAM_R), independent Rayleigh fading per axis (rayleigh_fading_SSD)
 *                                                     and erasures, GF(16) / GF(64) / GF(256); and its other switch settings
 *                                                     (APSK, plain QAM, rayleigh_fading, no erasure), pinned against
 *                                                     reference builds with those switches (oracle/Makefile CHAN_VARIANTS)
 *   ModelChannel_AWGN_256QAM_4D   channel.c:749-929   4-D rotated 16-QAM x 16-QAM, fading + erasures per axis, GF(256)
 * Parity status: PINNED -- tests/test_channel_oracle.py compares every LLR and the sorted (LLR, symbol) output of whole
 * frames with the REAL functions compiled into oracle/_ref, and tests/golden/channel_const.npz holds reference outputs
 * for the GPU box.  The raw constellation tables of include/channel.h are DATA of the fixture (dumped by
 * tests/golden/make_golden_channels.py); nothing of them is compiled in here.
 *
 * Split the way the product is split: orc_chan_* DRAW a frame's observations (harness side: RNG, modulation, fading,
 * erasures) and orc_const_llr turns them into the GF(q) LLR rows (the hot-path part the GPU kernel replaces). */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ems_oracle.h"

#define CH_PI 3.1415926536 /* channel.c:18 */

static float my_drand48f(orc_rng *r) { return (float)orc_drand48(r); } /* tools.c:71-73 */

/* channel.c:206-221 / :565-577 / :791-807: scale the table so that the average energy per 2 dimensions is one.
 * float accumulation in the reference's association: (t0*t0 + t1*t1) + acc, then (t2*t2 + t3*t3) + acc for D = 4. */
void orc_const_normalise(float *table, int q, int D)
{
    float nf = 0.0f;
    for (int i = 0; i < q; i++) {
        const float *t = table + (size_t)i * D;
        nf = t[0] * t[0] + t[1] * t[1] + nf;
        if (D == 4) nf = t[2] * t[2] + t[3] * t[3] + nf;
    }
    nf = (float)sqrt((double)((float)((D == 4 ? 2 : 1) * q) / nf));
    for (int i = 0; i < q * D; i++) table[i] = nf * table[i];
}

/* channel.c:347-358 / :398-409: rotation of a QAM table by alpha = 31.7 degrees (QAM_R) */
void orc_const_rotate(const float *qam, int q, float *out)
{
    const float alpha = (float)(31.7 * CH_PI / 180);
    for (int i = 0; i < q; i++) {
        out[2 * i + 0] = (float)((double)qam[2 * i + 0] * cos((double)alpha) - (double)qam[2 * i + 1] * sin((double)alpha));
        out[2 * i + 1] = (float)((double)qam[2 * i + 1] * cos((double)alpha) + (double)qam[2 * i + 0] * sin((double)alpha));
    }
}

/* channel.c:132-199: the 64-APSK table is written as radius*cos(PI*num/den), radius*sin(PI*num/den) in double, stored
 * as float; the (radius, num, den) triples are fixture data */
void orc_const_apsk64(const double *radius, const int *num, const int *den, float *out)
{
    for (int i = 0; i < 64; i++) {
        out[2 * i + 0] = (float)(radius[i] * cos(CH_PI * num[i] / den[i]));
        out[2 * i + 1] = (float)(radius[i] * sin(CH_PI * num[i] / den[i]));
    }
}

/* channel.c:361-375 (ModelChannel with the APSK switch, GF(256)): 8 rings of 32 points, float radius times the double cosine /
 * sine of angle*PI/16, stored as float; index = ring*32 + angle = the symbol's binary image.  The radii are fixture data. */
void orc_const_apsk256(const float *radius8, float *out)
{
    for (int ring = 0; ring < 8; ring++)
        for (int angle = 0; angle < 32; angle++) {
            out[2 * (ring * 32 + angle) + 0] = (float)((double)radius8[ring] * cos(angle * CH_PI / 16));
            out[2 * (ring * 32 + angle) + 1] = (float)((double)radius8[ring] * sin(angle * CH_PI / 16));
        }
}

/* channel.c:233 / :596 / :810: sigma from the SNR (no code rate here, unlike channel.c:51) */
float orc_chan_sigma(float EbN) { return (float)sqrt(1.0 / (2.0 * pow(10, (double)EbN / 10.0))); }

/* Observations of one frame.  images[n] = binary image of codeword symbol n (the reference's `som`, channel.c:249-252).
 * kind 0: ModelChannel_AWGN_64 (channel.c:255-266): per axis u, v; obs = point + noise; no attenuation (att := 1).
 * kind 1: ModelChannel, SSD + erasure (channel.c:636-670): per axis u, v, fading draw, erasure draw; the erased /
 *         rescaled attenuation multiplies the point.
 * kind 2: ModelChannel_AWGN_256QAM_4D (channel.c:829-856): per axis u, v, fading draw, the observation uses the RAW
 *         fading, then the erasure draw replaces the stored attenuation (the LLR sees the erased / rescaled one).
 * The compile-time variants of ModelChannel the reference ships switched off (channel.c:20-26):
 * kind 3: `rayleigh_fading` (channel.c:627-631): ONE fading draw per node before its axes, then per axis u, v; the same
 *         attenuation multiplies both coordinates and is stored for both.
 * kind 4: `rayleigh_fading_SSD` without `erasure` (channel.c:648-651): per axis u, v, fading draw; no erasure draw.
 * (no fading switch at all: kind 0 -- the point times the float 1.0 is the point.) */
void orc_chan_draw(orc_rng *r, int kind, const float *table, int D, int N, const int *images, float sigma, float *obs,
                   float *att)
{
    const float erasure_proba = (float)0.1; /* channel.c:593 / :771 */
    for (int n = 0; n < N; n++) {
        float node_a = 1.0f;
        if (kind == 3) node_a = (float)sqrt(-log((double)my_drand48f(r)));
        for (int d = 0; d < D; d++) {
            const float u = my_drand48f(r);
            const float v = my_drand48f(r);
            float a = 1.0f;
            const float pt = table[(size_t)images[n] * D + d];
            const double noise = (double)sigma * sqrt(-2.0 * log((double)u)) * cos(2.0 * CH_PI * (double)v);
            if (kind == 0) {
                obs[(size_t)n * D + d] = (float)((double)pt + noise);
            } else if (kind == 3) {
                a = node_a;
                obs[(size_t)n * D + d] = (float)((double)(pt * a) + noise);
            } else if (kind == 4) {
                a = (float)sqrt(-log((double)my_drand48f(r)));
                obs[(size_t)n * D + d] = (float)((double)(pt * a) + noise);
            } else {
                float rt = my_drand48f(r);
                a = (float)sqrt(-log((double)rt));
                if (kind == 2) obs[(size_t)n * D + d] = (float)((double)(pt * a) + noise);
                rt = my_drand48f(r);
                if (rt < erasure_proba) a = 0.0f;
                else a = (float)((double)a / sqrt((double)(1 - erasure_proba)));
                if (kind == 1) obs[(size_t)n * D + d] = (float)((double)(pt * a) + noise);
            }
            att[(size_t)n * D + d] = a;
        }
    }
}

/* The LLR rows (what the GPU kernel computes).  table [q][D] indexed by binary image; llr [N][q] indexed by symbol.
 * mode 0 (SUM, channel.c:277; with fading :681-684): D = 2, both quotients added in double, one rounding to float; with
 *         attenuations (att != NULL; the same on both axes in the reference) a node with attenuation <= 0 gets 0.
 * mode 1 (SSD, channel.c:690-700 / :870-886): axes with attenuation <= 0 are skipped, each quotient is added to the
 *         float running value in double and rounded back (the first assignment equals an addition to 0).
 * mode 2 (NEG, the older generation's max-is-best likelihood, EMS_HS_L-BubbleCheck_UBS_decoder_v2.c:1479):
 *         -(q0) - q1 in double, one rounding; 0 when the attenuation is <= 0. */
void orc_const_llr(const orc_gf *gf, const float *table, int q, int D, int mode, const float *obs, const float *att, int N,
                   float sigma, float *llr)
{
    const double den = 2.0 * (double)(sigma * sigma); /* 2.0*SQR(sigma): SQR in float (channel.h:13) */
    for (int n = 0; n < N; n++)
        for (int k = 0; k < q; k++) {
            const float *t = table + (size_t)gf->idx2bin[k] * D;
            const float *y = obs + (size_t)n * D;
            float tmp;
            if (mode == 0 || mode == 2) {
                const float f0 = att ? att[(size_t)n * D + 0] : 1.0f, f1 = att ? att[(size_t)n * D + 1] : 1.0f;
                const float a0 = y[0] - t[0] * f0, a1 = y[1] - f1 * t[1];
                if (!(f0 > 0)) tmp = 0.0f;
                else if (mode == 0) tmp = (float)((double)(a0 * a0) / den + (double)(a1 * a1) / den);
                else tmp = (float)(-((double)(a0 * a0) / den) - (double)(a1 * a1) / den);
            } else {
                tmp = 0.0f;
                for (int d = 0; d < D; d++) {
                    const float a = att[(size_t)n * D + d];
                    if (a > 0) {
                        const float e = y[d] - a * t[d];
                        tmp = (float)((double)tmp + (double)(e * e) / den);
                    }
                }
            }
            llr[(size_t)n * q + k] = tmp;
        }
}

/* ---- the older generation's channel front end (EMS_HS_L-BubbleCheck_UBS_decoder_v2.c:1167-1499) ---- */

/* v2.c:1184-1446: TABLEAU is the 16 x 16 grid (-7.5 + i/16, -7.5 + i%16), i = the symbol's binary image, scaled by
 * float R = 0.433860915637 (pinned through the likelihoods of all 256 symbols against the real function) */
void orc_v2_table(float *out /*[256][2]*/)
{
    const float R = (float)0.433860915637;
    for (int i = 0; i < 256; i++) {
        out[2 * i + 0] = R * (-7.5f + (float)(i / 16));
        out[2 * i + 1] = R * (-7.5f + (float)(i % 16));
    }
}

/* v2.c:1451-1470: per node one fading draw w, A = sqrt(-log(w)), then per axis u, v (drand48 straight into floats);
 * obs = point*A + noise, att = A on both axes.  sigma as orc_chan_sigma (v2.c:1451). */
void orc_v2_chan_draw(orc_rng *r, const float *table, int N, const int *images, float sigma, float *obs, float *att)
{
    for (int n = 0; n < N; n++) {
        const float w = my_drand48f(r);
        const float A = (float)sqrt(-log((double)w));
        for (int d = 0; d < 2; d++) {
            const float u = my_drand48f(r);
            const float v = my_drand48f(r);
            obs[2 * n + d] = (float)((double)(table[2 * images[n] + d] * A) + (double)sigma * sqrt(-2.0 * log((double)u)) * cos(2.0 * CH_PI * (double)v));
            att[2 * n + d] = A;
        }
    }
}

/* v2.c:1483-1497: q rounds of "first strict maximum above -1e5, then mark it -1e5": the likelihoods in decreasing
 * order with their symbols (decoder.likelihood / likelihood_index) */
void orc_v2_sort_likelihood(const float *like, int N, int q, float *s_like, int *s_idx)
{
    float *tmp = malloc(sizeof(float) * (size_t)q);
    for (int n = 0; n < N; n++) {
        memcpy(tmp, like + (size_t)n * q, sizeof(float) * (size_t)q);
        for (int k = 0; k < q; k++) {
            float best = -1e5f;
            int bg = -1;
            for (int g = 0; g < q; g++)
                if (tmp[g] > best) { best = tmp[g]; bg = g; }
            s_like[(size_t)n * q + k] = best;
            s_idx[(size_t)n * q + k] = bg;
            if (bg >= 0) tmp[bg] = -1e5f;
        }
    }
    free(tmp);
}
