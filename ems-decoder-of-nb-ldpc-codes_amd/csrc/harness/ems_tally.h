/*
 * ems_tally.h -- the bookkeeping of the Monte-Carlo loop (NB_LDPC.c:474-507) when the frames of one simulation are decoded by
 * several GPUs: which frames a device decodes in which round, what it contributes to the all-reduced counters, and the ordered
 * tally that prints the reference's progress line frame by frame and stops at exactly the 40th erroneous frame.
 *
 * Plain C, no GPU, no threads, no I/O of its own: ems_sim.c drives it with real decodes and an RCCL all-reduce,
 * tests/ctest/test_tally.c with a fake N-device decode and a barrier-based all-reduce.
 *
 * Sharding (same rule as harness.py): the stream of frames is cut in batches of `batch` frames; batch j goes to
 * device j mod ndev in round j / ndev.  Frames are independent, so there is no data-path exchange; per round the devices all-reduce
 * five 64-bit counters -- frames, erroneous frames, bit errors, undetected errors, sum of iterations (NB_LDPC.c:474-497) -- which is
 * what the global "40 erroneous frames" stop (NB_LDPC.c:506) is decided on, the same on every device.  The per-frame results stay in
 * host memory; device 0's thread walks them in frame order.
 */
#ifndef EMS_TALLY_H
#define EMS_TALLY_H

#ifdef __cplusplus
extern "C" {
#endif

enum { EMS_CNT_FRAMES = 0, EMS_CNT_ERR_FRAMES, EMS_CNT_BIT_ERRORS, EMS_CNT_UNDETECTED, EMS_CNT_SUM_IT, EMS_NCOUNTERS };

typedef struct {
    int nberr;   /* bit errors over the first K symbols (NB_LDPC.c:479-485) */
    int synd;    /* value Syndrom() returned (tools.c:284)                  */
    int iters;   /* iter + 1 (NB_LDPC.c:474)                                */
} ems_frame_result;

typedef struct {
    long long c[EMS_NCOUNTERS];
} ems_counters;

/* frames [*first, *first + *count) (0-based) of the batch device `dev` decodes in round `round`; *count = 0 when the stream has
 * run out (the device still takes part in the round's all-reduce, with zeros) */
static inline void ems_round_batch(long long n_frames, int batch, int ndev, long long round, int dev, long long *first, int *count)
{
    const long long j = round * ndev + dev, start = j * (long long)batch;
    long long n = n_frames - start;
    if (n > batch) n = batch;
    if (n < 0) n = 0;
    *first = start;
    *count = (int)n;
}

/* rounds needed for n_frames (every device runs this many) */
static inline long long ems_rounds_needed(long long n_frames, int batch, int ndev)
{
    const long long per_round = (long long)batch * ndev;
    return (n_frames + per_round - 1) / per_round;
}

/* one batch's contribution to the five counters: what a device hands to the all-reduce */
static inline void ems_batch_counters(const ems_frame_result *r, int count, long long out[EMS_NCOUNTERS])
{
    for (int i = 0; i < EMS_NCOUNTERS; i++) out[i] = 0;
    for (int b = 0; b < count; b++) {
        out[EMS_CNT_FRAMES] += 1;
        out[EMS_CNT_SUM_IT] += r[b].iters;
        out[EMS_CNT_BIT_ERRORS] += r[b].nberr;
        if (r[b].nberr != 0) {
            out[EMS_CNT_ERR_FRAMES] += 1;
            if (r[b].synd == 0) out[EMS_CNT_UNDETECTED] += 1;
        }
    }
}

/* The global stop, decided by every device on the same all-reduced numbers: `total` = running sum of the rounds' all-reduced
 * counters.  1 = no further round (the 40th erroneous frame lies in a round already decoded, or the stream is exhausted). */
static inline int ems_global_stop(const ems_counters *total, long long n_frames, int max_err)
{
    return total->c[EMS_CNT_ERR_FRAMES] >= max_err || total->c[EMS_CNT_FRAMES] >= n_frames;
}

/* Ordered tally of one round (NB_LDPC.c:474-507): res[d] / count[d] = device d's batch of this round, walked in device order =
 * frame order.  `line` (may be NULL) is called after every frame with the running counters -- the reference's progress line.
 * Stops at the frame that brings the erroneous frames to max_err: returns 1 then (frames behind it are dropped, like the frames
 * the reference never drew), else 0. */
static inline int ems_tally_round(ems_counters *run, const ems_frame_result *const *res, const int *count, int ndev, int max_err,
                                  void (*line)(const ems_counters *run, void *ctx), void *ctx)
{
    for (int d = 0; d < ndev; d++)
        for (int b = 0; b < count[d]; b++) {
            const ems_frame_result *r = &res[d][b];
            run->c[EMS_CNT_FRAMES] += 1;
            run->c[EMS_CNT_SUM_IT] += r->iters;
            run->c[EMS_CNT_BIT_ERRORS] += r->nberr;
            if (r->nberr != 0) {
                run->c[EMS_CNT_ERR_FRAMES] += 1;
                if (r->synd == 0) run->c[EMS_CNT_UNDETECTED] += 1;
            }
            if (line) line(run, ctx);
            if (run->c[EMS_CNT_ERR_FRAMES] == max_err) return 1;
        }
    return 0;
}

#ifdef __cplusplus
}
#endif
#endif /* EMS_TALLY_H */
