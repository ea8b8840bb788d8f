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

/* ---------------------------------------------------------------------------------------------------
 * Whole per-read pipeline on the host cores: index probe -> ungapped site filter -> DP + traceback, the same
 * control flow bbmap_amd/csrc/pipeline.hip runs on the device (AbstractMapThread.scoreNoIndels :762-856 and
 * the site filter of BBMapThread.scoreSlow :252-309).  One shared read-only index, one MSA per worker.
 * --------------------------------------------------------------------------------------------------- */
#include "index_oracle.h"

typedef struct {
    const orc_index *ix;
    const uint8_t *reads; int64_t n; int L;
    const int32_t *offsets, *keyScores; int nkeys;
    int maxColumns;
    volatile int64_t *next;
    int64_t mapped, dpJobs, cells;
} map_arg;

void orc_set_perfect(const uint8_t *bases, int blen, const uint8_t *ref, int reflen, int start, int stop, int32_t *out2);   /* rescue_oracle.c */

static uint8_t comp_base(uint8_t b) {
    switch (b) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; case 'N': return 'N'; default: return 0xFF; }
}

static void *map_worker(void *p) {
    map_arg *w = (map_arg *)p;
    const int L = w->L;
    orc_msa *m = orc_msa_new(((L + 31) / 32) * 32, w->maxColumns);
    orc_msa *mg = NULL;                       /* wider aligner for sites with gap arrays (gapped references), made on demand */
    const int gapColumns = w->maxColumns > 3000 ? w->maxColumns : 3000;   /* BBMap's maxColumns, as the GPU pipeline's gapped context */
    uint8_t *msg = NULL;
    uint8_t *bm = (uint8_t *)malloc((size_t)L), *ms = (uint8_t *)malloc((size_t)L + w->maxColumns + 64);
    int8_t *bs = (int8_t *)calloc((size_t)L, 1);
    const uint8_t *ref = w->ix->chromArr[1];
    const int reflen = w->ix->chromArrLen[1];
    const int maxSw = 70 + (L - 1) * 100, maxImp = maxSw - 495;
    const int minMsaLimit = -258 + (int)(0.56f * (float)maxSw);
    for (;;) {
        const int64_t i0 = __sync_fetch_and_add(w->next, 256);
        if (i0 >= w->n) break;
        const int64_t hi = (i0 + 256 < w->n) ? i0 + 256 : w->n;
        for (int64_t r = i0; r < hi; r++) {
            const uint8_t *bp = w->reads + r * L;
            for (int i = 0; i < L; i++) bm[i] = comp_base(bp[L - 1 - i]);
            orc_site sites[8];
            const int ns = orc_index_find(w->ix, bp, bm, L, bs, w->keyScores, w->offsets, w->nkeys, sites, 8, NULL);
            if (ns <= 0) continue;
            int near = 0, force = 0, sw[8];
            for (int s = 0; s < ns; s++) {
                if (sites[s].perfect) { sw[s] = maxSw; near++; }
                else {
                    const uint8_t *c = w->ix->chromArr[sites[s].chrom];
                    const uint8_t *bb = sites[s].strand ? bm : bp;
                    const int cl = w->ix->chromArrLen[sites[s].chrom], old = sites[s].score;
                    int32_t ps[2];
                    sw[s] = orc_score_no_indels(bb, L, c, cl, NULL, sites[s].start);
                    if (sw[s] < old && old >= maxImp && sites[s].stop - sites[s].start + 1 != L) {      /* AbstractMapThread.java:808-815 */
                        const int sw2 = orc_score_no_indels(bb, L, c, cl, NULL, sites[s].stop - L + 1);
                        if (sw2 >= maxImp) {
                            sw[s] = sw2; sites[s].start = sites[s].stop - L + 1;
                            orc_set_perfect(bb, L, c, cl, sites[s].start, sites[s].stop, ps); sites[s].perfect = ps[0]; sites[s].semiperfect = ps[1];
                        }
                    }
                    if (sw[s] >= maxImp) {
                        near++; sites[s].stop = sites[s].start + L - 1; sites[s].ngaps = 0;
                        if (sw[s] >= maxSw) sites[s].perfect = sites[s].semiperfect = 1;
                        else { orc_set_perfect(bb, L, c, cl, sites[s].start, sites[s].stop, ps); sites[s].perfect = ps[0]; sites[s].semiperfect = ps[1]; }
                    } else if (old >= maxImp) force = 1;
                }
            }
            int ok = near > 0;
            if ((force ? -near : near) < 1) {
                for (int s = 0; s < ns; s++) {
                    if (!(sw[s] < maxImp && !sites[s].semiperfect)) continue;
                    const uint8_t *bases = sites[s].strand ? bm : bp;
                    const uint8_t *c = w->ix->chromArr[sites[s].chrom];
                    const int clen = w->ix->chromArrLen[sites[s].chrom];
                    int32_t sc[8], mx[4];
                    if (sites[s].ngaps > 0) {             /* MSA.fillAndScoreLimited(..., gaps) + traceback(gapped) */
                        if (!mg) { mg = orc_msa_new(((L + 31) / 32) * 32, gapColumns); msg = (uint8_t *)malloc((size_t)L + gapColumns + 64 + 128 * 16); }
                        const int ga = sites[s].start - 4, gb = sites[s].stop + 4;
                        const int gmin = sw[s] > minMsaLimit ? sw[s] : minMsaLimit;
                        w->dpJobs++;
                        if (orc_fill_and_score_limited(mg, bases, L, c, clen, ga, gb, gmin, sites[s].gaps, sites[s].ngaps, sc, mx)) {
                            ok = 1;
                            orc_traceback(mg, bases, c, ga < 0 ? 0 : ga, gb > clen - 1 ? clen - 1 : gb, mx[0], mx[1], mx[2], 1, msg, L + gapColumns + 64 + 128 * 16);
                        }
                        continue;
                    }
                    int a = sites[s].start - 4, b = sites[s].stop + 4;
                    if (b - a + 1 > w->maxColumns) b = a + w->maxColumns - 1;
                    const int minscore = sw[s] > minMsaLimit ? sw[s] : minMsaLimit;
                    w->dpJobs++;
                    if (orc_fill_and_score_limited(m, bases, L, c, clen, a, b, minscore, NULL, 0, sc, mx)) {
                        ok = 1;
                        orc_traceback2(m, bases, c, a < 0 ? 0 : a, b > clen - 1 ? clen - 1 : b, mx[0], mx[1], mx[2], ms, L + w->maxColumns + 64);
                    }
                }
            }
            w->mapped += ok;
        }
    }
    (void)ref; (void)reflen;
    w->cells = m->iterationsLimited + m->iterationsUnlimited;
    if (mg) { w->cells += mg->iterationsLimited + mg->iterationsUnlimited; orc_msa_free(mg); free(msg); }
    free(bm); free(ms); free(bs);
    orc_msa_free(m);
    return NULL;
}

double orc_bench_map(const orc_index *ix, const uint8_t *reads, int64_t n, int L, const int32_t *offsets,
                     const int32_t *keyScores, int nkeys, int maxColumns, int threads,
                     int64_t *mapped, int64_t *dpJobs, int64_t *cells) {
    if (threads < 1) threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    map_arg *wa = (map_arg *)calloc((size_t)threads, sizeof(map_arg));
    volatile int64_t next = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        wa[t].ix = ix; wa[t].reads = reads; wa[t].n = n; wa[t].L = L; wa[t].offsets = offsets; wa[t].keyScores = keyScores;
        wa[t].nkeys = nkeys; wa[t].maxColumns = maxColumns; wa[t].next = &next;
        pthread_create(&th[t], NULL, map_worker, &wa[t]);
    }
    int64_t a = 0, b = 0, c = 0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); a += wa[t].mapped; b += wa[t].dpJobs; c += wa[t].cells; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (mapped) *mapped = a;
    if (dpJobs) *dpJobs = b;
    if (cells) *cells = c;
    free(th); free(wa);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
