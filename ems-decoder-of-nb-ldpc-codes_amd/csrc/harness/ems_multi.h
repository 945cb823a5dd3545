/*
 * ems_multi.h -- one Monte-Carlo simulation over several GPUs of one node, from C: one host thread per device, rounds of one batch
 * per device, per round ONE all-reduce of the five error counters (RCCL over xGMI in ems_sim.c) for the global 40-error stop, and
 * the ordered tally of ems_tally.h for the reference's progress line.  The decode and the all-reduce are callbacks, so that
 * tests/ctest/test_tally.c runs the very same loop with a fake N-device decode and a barrier-based all-reduce on the CPU.
 *
 * A device thread keeps the next round's batch in flight while it collects the current one (launch r+1, then collect r): host-side
 * frame generation and the tail of an early-exit batch overlap the next launch.  A round launched ahead of the stop is drained and
 * dropped, like a frame the reference never drew.
 *
 * The collective and the launch order (round 5).  The decode kernel is persistent (one workgroup per frame, resident until its frame is
 * done), so a collective kernel queued BEHIND round r+1's launch gets a CU only when workgroups of r+1 retire: at once in the usual
 * early-exit run, but not before r+1 ENDS when no frame of it converges (low SNR; fixed work) -- a host that waits for the all-reduce
 * of round r right after issuing it would then serialise the rounds.  Two ways out, both here:
 *   * allreduce_begin / allreduce_end (what ems_sim.c gives for RCCL): the all-reduce of round r is ISSUED right after round r is
 *     collected and WAITED FOR one round later, after round r+1 is collected -- by then it has certainly had its CU.  The global stop is
 *     therefore decided one round late (one more round is drained and dropped at the end of a simulation; the ordered tally never
 *     counts a frame behind the 40th erroneous one anyway).
 *   * allreduce alone (synchronous; a host-side sum under a barrier, as tests/ctest/test_tally.c and EMS_ALLREDUCE=host): nothing is
 *     queued on the device, the stop is decided in the round itself.
 */
#ifndef EMS_MULTI_H
#define EMS_MULTI_H

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ems_tally.h"

typedef struct {
    /* queue frames [first, first + count) on device `dev` in buffer slot `slot` (0 / 1); asynchronous; count > 0 */
    int (*launch)(void *ctx, int dev, int slot, long long first, int count);
    /* wait for that slot and fill res[0..count); res == NULL: drain only (the batch is dropped) */
    int (*collect)(void *ctx, int dev, int slot, int count, ems_frame_result *res);
    /* sum v[] over all devices; called once per round by every device thread, all of them with the same round.  Synchronous form;
     * used when allreduce_begin is NULL */
    int (*allreduce)(void *ctx, int dev, long long v[EMS_NCOUNTERS]);
    /* the progress line, called in frame order by device 0's thread (may be NULL) */
    void (*line)(const ems_counters *run, void *ctx);
    void *ctx;
    /* split form (both or neither): begin issues the all-reduce of v[] without waiting for it, end waits for the one begun last and
     * returns the sums in v[].  At most one is outstanding per device. */
    int (*allreduce_begin)(void *ctx, int dev, const long long v[EMS_NCOUNTERS]);
    int (*allreduce_end)(void *ctx, int dev, long long v[EMS_NCOUNTERS]);
} ems_multi_ops;

typedef struct {
    const ems_multi_ops *ops;
    int ndev, batch, max_err;
    long long n_frames;
    pthread_barrier_t bar;
    ems_frame_result **res;       /* [ndev] this round's results of each device */
    int *count;                   /* [ndev] */
    ems_counters run;             /* ordered tally (device 0's thread) */
    volatile int truncated;       /* the ordered tally reached max_err */
    volatile int failed;          /* some callback failed: everybody leaves after the current round */
} ems_multi_state;

typedef struct {
    ems_multi_state *st;
    int dev;
} ems_multi_arg;

static void *ems_multi_worker(void *argp)
{
    ems_multi_arg *a = (ems_multi_arg *)argp;
    ems_multi_state *st = a->st;
    const ems_multi_ops *ops = st->ops;
    const int dev = a->dev, ndev = st->ndev;
    const long long rounds = ems_rounds_needed(st->n_frames, st->batch, ndev);
    ems_counters total;                                   /* running sum of the all-reduced counters: identical on every device */
    memset(&total, 0, sizeof total);
    long long first;
    int cnt_cur = 0, cnt_next = 0, err = 0, pending = 0;
    const int lagged = ops->allreduce_begin != NULL && ops->allreduce_end != NULL;
    ems_round_batch(st->n_frames, st->batch, ndev, 0, dev, &first, &cnt_cur);
    if (cnt_cur > 0 && ops->launch(ops->ctx, dev, 0, first, cnt_cur)) err = 1;
    for (long long r = 0; r < rounds; r++) {
        const int slot = (int)(r & 1);
        cnt_next = 0;
        if (r + 1 < rounds) {                             /* the next round's batch goes out before this one is collected */
            ems_round_batch(st->n_frames, st->batch, ndev, r + 1, dev, &first, &cnt_next);
            if (!err && cnt_next > 0 && ops->launch(ops->ctx, dev, slot ^ 1, first, cnt_next)) err = 1;
        }
        if (!err && cnt_cur > 0 && ops->collect(ops->ctx, dev, slot, cnt_cur, st->res[dev])) err = 1;
        long long v[EMS_NCOUNTERS];
        ems_batch_counters(st->res[dev], err ? 0 : cnt_cur, v);
        if (err) st->failed = 1;
        if (lagged) {
            /* the all-reduce of round r - 1, issued a round ago, is waited for now that round r has been collected; then this round's
             * goes out.  (Every thread reaches both calls every round, whatever failed: its peers are inside the same collective.) */
            if (pending) {
                long long w[EMS_NCOUNTERS];
                if (ops->allreduce_end(ops->ctx, dev, w)) { st->failed = 1; err = 1; memset(w, 0, sizeof w); }
                for (int i = 0; i < EMS_NCOUNTERS; i++) total.c[i] += w[i];
            }
            if (ops->allreduce_begin(ops->ctx, dev, v)) { st->failed = 1; err = 1; }
            pending = 1;
        } else {
            if (ops->allreduce(ops->ctx, dev, v)) { st->failed = 1; err = 1; }    /* (every thread still reaches the barriers below) */
            for (int i = 0; i < EMS_NCOUNTERS; i++) total.c[i] += v[i];
        }
        st->count[dev] = err ? 0 : cnt_cur;
        pthread_barrier_wait(&st->bar);                   /* every device's results of the round are in place */
        /* The failure flag is LATCHED here, between the two barriers: every write of round r happened before the first barrier and
         * no thread can be past the second one, so all threads read the same value and take the same decision for round r.  (Read
         * after the second barrier, a fast thread already failing in round r + 1 could make a slow one leave at r while the fast
         * one waits for it in r + 1's all-reduce or barrier for ever.) */
        const int failed_now = st->failed;
        if (dev == 0 && !failed_now && !st->truncated)        /* (lagged stop: a round behind the truncation is never tallied) */
            st->truncated = ems_tally_round(&st->run, (const ems_frame_result *const *)st->res, st->count, ndev, st->max_err, ops->line, ops->ctx);
        pthread_barrier_wait(&st->bar);                   /* the tally has read them */
        const int stop = failed_now || ems_global_stop(&total, st->n_frames, st->max_err);
        if (stop) {
            if (cnt_next > 0 && !err) (void)ops->collect(ops->ctx, dev, slot ^ 1, cnt_next, NULL);   /* launched ahead of the stop: dropped */
            break;
        }
        cnt_cur = cnt_next;
    }
    if (pending) {                                        /* nothing is left outstanding on the collective's stream */
        long long w[EMS_NCOUNTERS];
        if (ops->allreduce_end(ops->ctx, dev, w)) st->failed = 1;
    }
    return NULL;
}

/* Runs the simulation; *final receives the ordered tally (what the reference's counters hold when its loop ends).  Returns 0, or
 * -1 when a callback failed or threads could not be started. */
static inline int ems_multi_run(const ems_multi_ops *ops, int ndev, long long n_frames, int batch, int max_err, ems_counters *final)
{
    if (!ops || ndev < 1 || batch < 1 || n_frames < 0) return -1;
    ems_multi_state st;
    memset(&st, 0, sizeof st);
    st.ops = ops; st.ndev = ndev; st.batch = batch; st.max_err = max_err; st.n_frames = n_frames;
    st.res = (ems_frame_result **)calloc((size_t)ndev, sizeof *st.res);
    st.count = (int *)calloc((size_t)ndev, sizeof *st.count);
    pthread_t *th = (pthread_t *)calloc((size_t)ndev, sizeof *th);
    ems_multi_arg *args = (ems_multi_arg *)calloc((size_t)ndev, sizeof *args);
    int rc = (st.res && st.count && th && args) ? 0 : -1;
    for (int d = 0; d < ndev && !rc; d++)
        if (!(st.res[d] = (ems_frame_result *)calloc((size_t)batch, sizeof **st.res))) rc = -1;
    if (!rc && pthread_barrier_init(&st.bar, NULL, (unsigned)ndev)) rc = -1;
    if (!rc) {
        int started = 0;
        for (int d = 0; d < ndev; d++) {
            args[d].st = &st; args[d].dev = d;
            if (pthread_create(&th[d], NULL, ems_multi_worker, &args[d])) break;
            started++;
        }
        if (started != ndev) {                            /* a barrier built for ndev threads cannot be left by fewer: give up loudly */
            fprintf(stderr, "ems_multi_run: could not start %d device threads\n", ndev);
            exit(EXIT_FAILURE);
        }
        for (int d = 0; d < started; d++) pthread_join(th[d], NULL);
        pthread_barrier_destroy(&st.bar);
        if (st.failed) rc = -1;
    }
    if (final) *final = st.run;
    if (st.res)
        for (int d = 0; d < ndev; d++) free(st.res[d]);
    free(st.res); free(st.count); free(th); free(args);
    return rc;
}

#endif /* EMS_MULTI_H */
