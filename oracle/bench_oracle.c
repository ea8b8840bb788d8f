/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 * Multi-threaded timing harness for the CPU restatement: bench.py's cpu_baseline leg.
 * Mirrors how the reference runs the DP on a host: one worker per core, each with a private
 * MSA instance (current/align2/AbstractMapThread.java:133-134), jobs pulled from a shared list.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "msa11ts_oracle.h"

typedef struct {
    int64_t read_off, ref_off;
    int32_t read_len, ref_len, refStartLoc, refEndLoc, minScore, flags;
} orc_job;   /* same layout as bbmsa_job (include/bbmap_amd.h) */

typedef struct {
    const orc_job *jobs; int64_t n;
    const uint8_t *reads, *refs;
    int maxRows, maxColumns;
    volatile int64_t *next;
    int64_t cells;            /* out: iterations */
    int64_t checksum;         /* out: keeps the work live */
    int traceback;
} worker_arg;

static void *worker(void *p) {
    worker_arg *w = (worker_arg *)p;
    orc_msa *m = orc_msa_new(w->maxRows, w->maxColumns);
    uint8_t *ms = (uint8_t *)malloc((size_t)w->maxRows + w->maxColumns + 64);
    int64_t sum = 0;
    for (;;) {
        const int64_t i = __sync_fetch_and_add(w->next, 64);
        if (i >= w->n) break;
        const int64_t hi = (i + 64 < w->n) ? i + 64 : w->n;
        for (int64_t k = i; k < hi; k++) {
            const orc_job *j = &w->jobs[k];
            int32_t sc[8], mx[4];
            const uint8_t *rd = w->reads + j->read_off, *rf = w->refs + j->ref_off;
            const int n = orc_fill_and_score_limited(m, rd, j->read_len, rf, j->ref_len,
                                                     j->refStartLoc, j->refEndLoc, j->minScore, NULL, 0, sc, mx);
            if (n) {
                sum += sc[0] + sc[1];
                if (w->traceback) {
                    const int a = j->refStartLoc < 0 ? 0 : j->refStartLoc;
                    const int b = j->refEndLoc > j->ref_len - 1 ? j->ref_len - 1 : j->refEndLoc;
                    const int L = orc_traceback2(m, rd, rf, a, b, mx[0], mx[1], mx[2], ms, w->maxRows + w->maxColumns + 64);
                    sum += L;
                }
            }
        }
    }
    w->cells = m->iterationsLimited + m->iterationsUnlimited;
    w->checksum = sum;
    free(ms);
    orc_msa_free(m);
    return NULL;
}

/* Runs fillAndScoreLimited (+traceback) over jobs[0..n) on `threads` workers.
 * Returns elapsed seconds; *cells / *checksum are totals. */
double orc_bench_align(const orc_job *jobs, int64_t n, const uint8_t *reads, const uint8_t *refs,
                       int maxRows, int maxColumns, int threads, int traceback,
                       int64_t *cells, int64_t *checksum) {
    if (threads < 1) threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    worker_arg *wa = (worker_arg *)calloc((size_t)threads, sizeof(worker_arg));
    volatile int64_t next = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        wa[t].jobs = jobs; wa[t].n = n; wa[t].reads = reads; wa[t].refs = refs;
        wa[t].maxRows = maxRows; wa[t].maxColumns = maxColumns; wa[t].next = &next; wa[t].traceback = traceback;
        pthread_create(&th[t], NULL, worker, &wa[t]);
    }
    int64_t c = 0, s = 0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); c += wa[t].cells; s += wa[t].checksum; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (cells) *cells = c;
    if (checksum) *checksum = s;
    free(th); free(wa);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
