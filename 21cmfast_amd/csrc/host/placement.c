/* Placement of work spectra in HBM (round 5): see the comment of c21_place_work_partner. */
#include <stdio.h>
#include <stdlib.h>

#include "c21cm_grid.h"
#include "../hip/c21hip.h"

/* Placement of the second work spectrum of a two-grid launch (round 5, late).  Two buffers that one launch
 * WRITES (pass X's outputs, pass Y in place) must not sit in the same physical region of the HBM: a pair from one
 * region costs pass Y 17-21 % at 1024^3 and 10-13 % at 512^3 (pair matrices in profiles/r05_placement_study.txt:
 * classes contiguous in allocation order; on an empty GPU the other class starts 77-136 GB into a linear walk, on
 * a used one after 32 GB; reads do not care, neither do single-grid launches).  hipMalloc carves consecutive
 * allocations out of one region, so the default is the slow pair more often than not -- the "two speeds" of the
 * line passes since round 3.  The class turned out to be the parity of the 32 GiB stripe of PHYSICAL memory a
 * buffer lies in (study, section 12); virtual addresses do not tell it; a timed launch does:
 *   phase 1: chunks of 16 GB (at least the buffer's size) are allocated one after the other and all held; the head
 *            of each is timed with the partner in the two-grid pass Y (c21hip_probe_pass_y2) until two chunks
 *            differ by 8 % (the classes are 13-17 % apart): the faster one marks a good region;
 *   phase 2: that chunk alone is freed and buffers of the exact size are allocated (and held) until one times as
 *            fast: small requests are served from the small holes first -- next to the partner, as a rule -- and
 *            from the freed chunk once those are filled;
 *   then everything but the chosen buffer is freed and the workspace slot adopts it.
 * Up to C21CM_WS_PLACE_GB (default 200) GB are held for some milliseconds, three quarters of what is free at most;
 * nothing found: a plain allocation.  Once per slot and size (the workspace keeps the buffer); C21CM_WS_PLACE=0:
 * plain allocation; C21CM_WS_TRACE=1 prints the candidates. */
float *c21_place_work_partner(int slot_partner, int slot_new, size_t bytes, int nx, int ny, int nz, void *stream) {
    size_t have = 0;
    float *cur = (float *)c21hip_ws_peek(slot_new, &have);
    if (cur && have >= bytes) return cur; /* placed before */
    static int on = -1;
    static double max_gb = 200.;
    if (on < 0) {
        const char *e = getenv("C21CM_WS_PLACE"), *g = getenv("C21CM_WS_PLACE_GB");
        on = (e && e[0] == '0') ? 0 : 1;
        if (g && atof(g) > 0.) max_gb = atof(g);
    }
    float *partner = (float *)c21hip_ws_peek(slot_partner, &have);
    if (!on || !partner || have < bytes || bytes < ((size_t)256 << 20)) return (float *)c21hip_ws(slot_new, bytes);
    enum { MAXH = 128 };
    void *held[MAXH];
    int n_held = 0;
    size_t budget = (size_t)(max_gb * 1073741824.);
    {
        const size_t fr = c21hip_free_bytes() / 4 * 3;
        if (fr < budget) budget = fr;
    }
    const int trace = getenv("C21CM_WS_TRACE") != NULL;
    const int reps = 3;
    size_t chunk = (size_t)16 << 30;
    if (chunk > budget / 4) chunk = budget / 4; /* (a short walk, e.g. under pytest-xdist: smaller steps) */
    if (chunk < bytes) chunk = bytes;
    size_t used = 0;
    /* ---- phase 1 */
    int i_best = -1, i_worst = -1;
    float t_of[MAXH];
    while (n_held < MAXH / 2 && used + chunk <= budget) {
        void *p = c21hip_raw_alloc(chunk);
        if (!p) break;
        used += chunk;
        held[n_held] = p;
        float t = 0.f;
        const int st = c21hip_probe_pass_y2(partner, (float *)p, nx, ny, nz, reps, &t, stream);
        t_of[n_held] = (st || !(t > 0.f)) ? -1.f : t;
        n_held++;
        if (t_of[n_held - 1] < 0.f) break;
        if (trace)
            fprintf(stderr, "[place] slot %d chunk %d (%.0f GB in): %.4f ms\n", slot_new, n_held - 1, used / 1073741824., t);
        if (i_best < 0 || t < t_of[i_best]) i_best = n_held - 1;
        if (i_worst < 0 || t > t_of[i_worst]) i_worst = n_held - 1;
        if (t_of[i_worst] > 1.08f * t_of[i_best]) break; /* both classes seen */
    }
    void *chosen = NULL;
    if (i_best >= 0 && i_worst >= 0 && t_of[i_worst] > 1.08f * t_of[i_best]) {
        const float t_good = t_of[i_best];
        if (chunk == bytes) { /* (boxes whose spectra are chunk-sized: the chunk is the buffer) */
            chosen = held[i_best];
            held[i_best] = NULL;
        } else {
            /* ---- phase 2 */
            c21hip_raw_free(held[i_best]);
            held[i_best] = NULL;
            while (n_held < MAXH) {
                void *p = c21hip_raw_alloc(bytes);
                if (!p) break;
                float t = 0.f;
                const int st = c21hip_probe_pass_y2(partner, (float *)p, nx, ny, nz, reps, &t, stream);
                if (trace) fprintf(stderr, "[place] slot %d exact-size candidate: %.4f ms\n", slot_new, t);
                if (!st && t > 0.f && t < 1.04f * t_good) {
                    chosen = p;
                    break;
                }
                held[n_held++] = p;
                if (st) break;
            }
        }
    }
    for (int i = 0; i < n_held; i++)
        if (held[i]) c21hip_raw_free(held[i]);
    if (!chosen) return (float *)c21hip_ws(slot_new, bytes);
    if (c21hip_ws_adopt(slot_new, chosen, bytes)) {
        c21hip_raw_free(chosen);
        return (float *)c21hip_ws(slot_new, bytes);
    }
    return (float *)chosen;
}
