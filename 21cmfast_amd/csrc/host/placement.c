/* Placement of work spectra in HBM (round 5): see the comment of c21_place_work_partner. */
#include <stdio.h>
#include <stdlib.h>

#include "c21cm_grid.h"
#include "../hip/c21hip.h"

/* Placement of the second work spectrum of a two-grid launch (round 5, late).  Two buffers that one launch
 * WRITES (pass X's outputs, pass Y in place) must not sit in the same physical region of the HBM: a pair from one
 * region costs pass Y 17-21 % at 1024^3 and 10 % at 512^3 (pair matrices in profiles/r05_placement_study.txt: two
 * classes, contiguous in allocation order, one boundary per ~35 GB; reads do not care, neither do single-grid
 * launches).  hipMalloc carves consecutive allocations out of one region, so the default is the slow pair more
 * often than not -- the "two speeds" of the line passes.  Addresses do not tell the region; a timed launch does:
 * candidates are allocated one after the other (rejected ones stay allocated as spacers until the end, with 4 GB
 * spacers in between for small boxes), each is timed with the partner in the two-grid pass Y, and as soon as two
 * candidates differ by 8 % (the classes are 13-17 % apart) the faster one is taken.  Up to C21CM_WS_PLACE_GB (default 48) of free memory is
 * walked, a quarter of what is free at most; nothing found: the first candidate.  Once per slot and size (the
 * workspace keeps the buffer); C21CM_WS_PLACE=0: plain allocation. */
float *c21_place_work_partner(int slot_partner, int slot_new, size_t bytes, int nx, int ny, int nz, void *stream) {
    size_t have = 0;
    float *cur = (float *)c21hip_ws_peek(slot_new, &have);
    if (cur && have >= bytes) return cur; /* placed before */
    static int on = -1;
    static double max_gb = 48.;
    if (on < 0) {
        const char *e = getenv("C21CM_WS_PLACE"), *g = getenv("C21CM_WS_PLACE_GB");
        on = (e && e[0] == '0') ? 0 : ((e && e[0] == '2') ? 2 : 1); /* 2: walk the whole budget, take the fastest */
        if (g && atof(g) > 0.) max_gb = atof(g);
    }
    float *partner = (float *)c21hip_ws_peek(slot_partner, &have);
    if (!on || !partner || have < bytes || bytes < ((size_t)256 << 20)) return (float *)c21hip_ws(slot_new, bytes);
    enum { MAXC = 96 };
    void *held[MAXC];
    int n_held = 0;
    size_t budget = (size_t)(max_gb * 1073741824.);
    {
        const size_t fr = c21hip_free_bytes() / 4;
        if (fr < budget) budget = fr;
    }
    const size_t spacer = (bytes < ((size_t)2 << 30)) ? ((size_t)4 << 30) : 0;
    size_t used = 0;
    void *first = NULL, *chosen = NULL, *best = NULL;
    float t_slowest = 0.f, t_first = 0.f, t_best = 0.f;
    const int trace = getenv("C21CM_WS_TRACE") != NULL;
    const int reps = 3;
    while (n_held + 2 <= MAXC && used + bytes <= budget) {
        void *p = c21hip_raw_alloc(bytes);
        if (!p) break;
        used += bytes;
        float t = 0.f;
        if (c21hip_probe_pass_y2(partner, (float *)p, nx, ny, nz, reps, &t, stream) || !(t > 0.f)) {
            if (!first) first = p;
            else held[n_held++] = p;
            break;
        }
        if (!first) first = p, t_first = t;
        else held[n_held++] = p;
        if (t > t_slowest) t_slowest = t;
        if (!best || t < t_best) best = p, t_best = t;
        if (trace) fprintf(stderr, "[place] slot %d candidate %p after %.1f GB: %.4f ms\n", slot_new, p, used / 1073741824., t);
        if (on == 2) goto next;
        if (t_slowest > 1.08f * t_best) { /* both classes seen: the fastest so far */
            chosen = best;
            break;
        }
    next:
        if (spacer && used + spacer + bytes <= budget) {
            void *sp = c21hip_raw_alloc(spacer);
            if (!sp) break;
            held[n_held++] = sp;
            used += spacer;
        }
    }
    if (!chosen) chosen = (on == 2 && best) ? best : first;
    (void)t_first;
    if (first && first != chosen) c21hip_raw_free(first);
    for (int i = 0; i < n_held; i++)
        if (held[i] != chosen) c21hip_raw_free(held[i]);
    if (!chosen) return (float *)c21hip_ws(slot_new, bytes);
    if (c21hip_ws_adopt(slot_new, chosen, bytes)) {
        c21hip_raw_free(chosen);
        return (float *)c21hip_ws(slot_new, bytes);
    }
    return (float *)chosen;
}
