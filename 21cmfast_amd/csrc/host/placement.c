/* Placement of work spectra in HBM (round 5; made safe for shared devices in round 6): see c21_place_work_partner. */
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/file.h>
#include <time.h>
#include <unistd.h>

#include "c21cm_grid.h"
#include "../hip/c21hip.h"

/* Placement of the second work spectrum of a two-grid launch.  Two buffers that one launch WRITES (pass X's
 * outputs, pass Y in place) must not sit in 32 GiB stripes of PHYSICAL memory of equal parity: such a pair costs
 * pass Y 17-21 % at 1024^3 and 10-13 % at 512^3 (profiles/r05_placement_study.txt, sections 1, 11, 12; reads do not
 * care, neither do single-grid launches).  hipMalloc carves consecutive allocations out of one stripe, so the
 * default is the slow pair more often than not -- the "two speeds" of the line passes since round 3.  Virtual
 * addresses do not tell the stripe; a timed launch does:
 *   phase 1: chunks of 16 GB (at least the buffer's size) are allocated one after the other and all held; the head
 *            of each is timed with the partner in the two-grid pass Y (c21hip_probe_pass_y2) until two chunks
 *            differ by 8 % (the classes are 13-17 % apart): the faster one marks a good region;
 *   phase 2: that chunk alone is freed and buffers of the exact size are allocated (and held) until one times as
 *            fast: small requests are served from the small holes first -- next to the partner, as a rule -- and
 *            from the freed chunk once those are filled;
 *   then everything but the chosen buffer is freed and the workspace slot adopts it.
 *
 * The walk holds memory it does not need (for some tens of milliseconds), so it only runs where that cannot hurt
 * anybody (round 6):
 *   - only when THIS process is the only one holding memory on the device (the KFD's per-process accounting,
 *     c21hip_device_tenants); another tenant -- a second pytest-xdist worker, a second bench.py, somebody's
 *     notebook -- means a plain allocation.  Where the accounting cannot be read the walk is bounded to four
 *     buffers and skipped if more than half of the device is in use;
 *   - one walker at a time per device (a non-blocking flock; the loser allocates plainly);
 *   - everything held, in BOTH phases, counts against the budget: C21CM_WS_PLACE_GB (default 200) GB and three
 *     quarters of what is free at most (phase 2 used to run on until hipMalloc failed);
 *   - a walk that found nothing is not repeated for that slot and size; a decision stands while the slot and its
 *     partner hold the buffers it was taken for, and is retaken when either changed (a slot first allocated plainly,
 *     or placed against another partner, is placed then).
 *   - a time budget: C21CM_WS_PLACE_MS per walk (default 300 ms), five times that per process.
 * And it is OPT-IN (round 6, late): what a walk costs is erratic on this driver -- an allocation of 16 GB returns in
 * 0.1 ms or in 2 s, whichever API hands it out (the memory is cleared when it was dirty) -- and a config-5
 * evolution whose ComputeIonizedBox calls take 3.3 s in all spent another 2-14 s walking for its four x_e
 * spectra (profiles/r06_placement_vmm.txt section 5): a drop-in library does not take that decision for its
 * caller.  C21CM_WS_PLACE=1 or c21cm_placement_set(1) turn it on (steady-state throughput runs: bench.py does, and
 * says so in its line), =force / 2 also with other tenants; C21CM_WS_TRACE=1 prints the candidates;
 * c21cm_placement_report() tells what the last call decided and what it cost. */

enum { PLACE_SLOTS = 384, MAXH = 128, PLACE_VMM_DEFAULT = 1 };
enum {
    PL_PLACED = 0,       /* a faster region was found and adopted */
    PL_OFF = 1,          /* C21CM_WS_PLACE=0 / no partner / small buffer */
    PL_TENANTS = 2,      /* another process holds memory on the device */
    PL_LOCKED = 3,       /* another process is walking */
    PL_NOTHING = 4,      /* no candidate differed within the budget */
    PL_REMEMBERED = 5,   /* an earlier walk for this slot and size found nothing */
    PL_BUSY_DEVICE = 6,  /* tenancy unknown and more than half of the device in use */
    PL_TIME = 7,         /* the walk's time budget ran out (this walk's, or the process') */
};
typedef struct {
    size_t bytes;
    void *partner, *ptr;
    int decided, mode_gen;
    size_t failed_bytes; /* a walk for this size found nothing */
} place_rec;
static place_rec g_rec[PLACE_SLOTS];
static struct {
    int outcome, probes, tenants, walks, slot;
    double held_gb, wall_ms;
    float chosen_ms, first_ms;
} g_last = {-1, 0, -1, 0, -1, 0., 0., 0.f, 0.f};
static double g_walk_ms_total; /* wall time this process has spent walking */

static double wall_ms_now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return 1e3 * (double)t.tv_sec + 1e-6 * (double)t.tv_nsec;
}

/* {outcome, GB held at the peak, timed probes, ms of the chosen pair, ms of the first candidate, wall ms of the
 * decision, tenants seen (-1 unknown), walks so far} of the last c21_place_work_partner call that had to decide */
int c21cm_placement_report(double out[8]) {
    if (!out || g_last.outcome < 0) return C21CM_VALUE_ERROR;
    out[0] = g_last.outcome, out[1] = g_last.held_gb, out[2] = g_last.probes, out[3] = g_last.chosen_ms;
    out[4] = g_last.first_ms, out[5] = g_last.wall_ms, out[6] = g_last.tenants, out[7] = g_last.walks;
    return 0;
}

/* a candidate region: hipMalloc'ed, or (C21CM_WS_PLACE_ALLOC=vmm) physical memory from hipMemCreate with only the
 * head the probe touches mapped -- hipMalloc pays 38-45 ms per GB on this driver (profiles/r06_placement_vmm.txt section 1), the
 * walk's 16 GB chunks cost 0.6 s each that way */
typedef struct {
    void *ptr, *vmm;
} cand;
static int cand_alloc(cand *c, size_t phys_bytes, size_t map_bytes, int vmm) {
    c->ptr = c->vmm = NULL;
    if (vmm) {
        c->vmm = c21hip_vmm_chunk(phys_bytes, map_bytes, &c->ptr);
        return c->vmm ? 0 : 1;
    }
    c->ptr = c21hip_raw_alloc(phys_bytes);
    return c->ptr ? 0 : 1;
}
static void cand_free(cand *c) {
    if (c->vmm) c21hip_vmm_chunk_free(c->vmm);
    else if (c->ptr) c21hip_raw_free(c->ptr);
    c->ptr = c->vmm = NULL;
}

/* Whether the walk runs at all.  OFF unless asked for (round 6): c21cm_placement_set(1) / C21CM_WS_PLACE=1 turn it
 * on, 2 / "force" also with other tenants on the device, 0 off; -1 back to the environment.  A change of the
 * setting re-opens the decisions taken under the old one. */
static int g_mode = -1, g_mode_gen = 0;
int c21cm_placement_set(int mode) {
    if (mode < -1 || mode > 2) return C21CM_VALUE_ERROR;
    if (mode != g_mode) g_mode_gen++;
    g_mode = mode;
    return 0;
}
static int placement_mode(void) {
    if (g_mode >= 0) return g_mode;
    const char *e = getenv("C21CM_WS_PLACE");
    if (!e || !e[0] || e[0] == '0') return 0;
    return e[0] == 'f' ? 2 : 1;
}

static float *plain(place_rec *rec, int slot_new, size_t bytes, void *partner, int outcome, double t0) {
    float *p = (float *)c21hip_ws(slot_new, bytes); /* (keeps a buffer the slot already holds) */
    if (rec) {
        rec->bytes = bytes, rec->partner = partner, rec->ptr = p, rec->decided = 1, rec->mode_gen = g_mode_gen;
    }
    g_last.outcome = outcome, g_last.slot = slot_new;
    g_last.wall_ms = wall_ms_now() - t0;
    return p;
}

float *c21_place_work_partner(int slot_partner, int slot_new, size_t bytes, int nx, int ny, int nz, void *stream) {
    size_t have = 0, phave = 0;
    float *cur = (float *)c21hip_ws_peek(slot_new, &have);
    float *partner = (float *)c21hip_ws_peek(slot_partner, &phave);
    place_rec *rec = (slot_new >= 0 && slot_new < PLACE_SLOTS) ? &g_rec[slot_new] : NULL;
    if (cur && have >= bytes &&
        (!rec || (rec->decided && rec->mode_gen == g_mode_gen && rec->ptr == (void *)cur && rec->partner == (void *)partner)))
        return cur; /* the decision taken for this pair of buffers stands */
    const double t0 = wall_ms_now();
    const char *g = getenv("C21CM_WS_PLACE_GB");
    const int off = placement_mode() == 0, force = placement_mode() == 2;
    const double max_gb = (g && atof(g) > 0.) ? atof(g) : 200.;
    g_last.probes = 0, g_last.held_gb = 0., g_last.chosen_ms = g_last.first_ms = 0.f, g_last.tenants = -1;
    if (off || !partner || phave < bytes || bytes < ((size_t)256 << 20))
        return plain(rec, slot_new, bytes, partner, PL_OFF, t0);
    if (rec && rec->failed_bytes == bytes && !force) return plain(rec, slot_new, bytes, partner, PL_REMEMBERED, t0);

    const int tenants = c21hip_device_tenants();
    g_last.tenants = tenants;
    if (tenants > 1 && !force) return plain(rec, slot_new, bytes, partner, PL_TENANTS, t0);
    const size_t free_now = c21hip_free_bytes(), total = c21hip_total_bytes();
    size_t budget = (size_t)(max_gb * 1073741824.);
    if (free_now / 4 * 3 < budget) budget = free_now / 4 * 3;
    if (tenants < 0 && !force) { /* nobody can tell who else is here: stay small, stay away from a busy device */
        if (total && total - free_now > total / 2) return plain(rec, slot_new, bytes, partner, PL_BUSY_DEVICE, t0);
        if (budget > 4 * bytes) budget = 4 * bytes;
    }
    /* one walker per device at a time */
    char lock_path[64];
    snprintf(lock_path, sizeof(lock_path), "/tmp/c21cm_place_dev%d.lock", c21hip_current_device());
    const int lock_fd = open(lock_path, O_CREAT | O_RDWR, 0666);
    if (lock_fd >= 0 && flock(lock_fd, LOCK_EX | LOCK_NB) != 0) {
        close(lock_fd);
        return plain(rec, slot_new, bytes, partner, PL_LOCKED, t0);
    }

    /* Time budget (round 6): the walk buys 1-4 ms per 512^3 call, and what it costs is erratic -- the driver
     * clears memory it hands out when it was dirty, 38-45 ms per GB: a config-5 evolution spent 16 of its 21 s
     * walking before this (profiles/r06_placement_vmm.txt section 5).  C21CM_WS_PLACE_MS per walk (default 300),
     * five times that per process; phase 1 takes its chunks from hipMemCreate with only the probed head mapped
     * (no clearing, ~0.1 ms per chunk), only the exact-size candidates of phase 2 come from hipMalloc. */
    const char *etm = getenv("C21CM_WS_PLACE_MS");
    const double walk_ms = (etm && atof(etm) > 0.) ? atof(etm) : 300.;
    if (g_walk_ms_total > 5. * walk_ms && !force) {
        if (lock_fd >= 0) {
            (void)flock(lock_fd, LOCK_UN);
            close(lock_fd);
        }
        return plain(rec, slot_new, bytes, partner, PL_TIME, t0);
    }
    const double deadline = t0 + walk_ms;
    int timed_out = 0;
    cand held[MAXH];
    float t_of[MAXH];
    const char *ea = getenv("C21CM_WS_PLACE_ALLOC");
    const int vmm = ea ? (ea[0] == 'v') : PLACE_VMM_DEFAULT;
    int n_held = 0;
    const int trace = getenv("C21CM_WS_TRACE") != NULL;
    const int reps = 3;
    size_t chunk = (size_t)16 << 30;
    if (chunk > budget / 4) chunk = budget / 4; /* (a short walk: smaller steps) */
    if (chunk < bytes) chunk = bytes;
    size_t used = 0, peak = 0;
    int i_best = -1, i_worst = -1;
    float t_cur = -1.f;
    g_last.walks++;
    /* a buffer the slot already holds (allocated plainly earlier, or placed against another partner) is the
     * first candidate: if it is of the fast class nothing has to move */
    if (cur && have >= bytes) {
        float t = 0.f;
        if (!c21hip_probe_pass_y2(partner, cur, nx, ny, nz, reps, &t, stream) && t > 0.f) t_cur = t;
        g_last.probes++;
        if (trace) fprintf(stderr, "[place] slot %d current buffer: %.4f ms\n", slot_new, t);
    }
    /* ---- phase 1 */
    while (n_held < MAXH / 2 && used + chunk <= budget) {
        if (cand_alloc(&held[n_held], chunk, bytes, vmm)) break;
        used += chunk;
        if (used > peak) peak = used;
        float t = 0.f;
        const int st = c21hip_probe_pass_y2(partner, (float *)held[n_held].ptr, nx, ny, nz, reps, &t, stream);
        g_last.probes++;
        t_of[n_held] = (st || !(t > 0.f)) ? -1.f : t;
        n_held++;
        if (t_of[n_held - 1] < 0.f) break;
        if (trace)
            fprintf(stderr, "[place] slot %d chunk %d (%.0f GB in): %.4f ms\n", slot_new, n_held - 1, used / 1073741824., t);
        if (n_held == 1) g_last.first_ms = t;
        if (i_best < 0 || t < t_of[i_best]) i_best = n_held - 1;
        if (i_worst < 0 || t > t_of[i_worst]) i_worst = n_held - 1;
        if (t_cur > 0.f && t_cur < 0.96f * t) break;                /* the current buffer is of the fast class */
        if (t_of[i_worst] > 1.08f * t_of[i_best]) break;            /* both classes seen */
        if (wall_ms_now() > deadline) {
            timed_out = 1;
            break;
        }
    }
    cand chosen = {NULL, NULL};
    int keep_current = 0;
    if (t_cur > 0.f && i_worst >= 0 && t_cur < 0.96f * t_of[i_worst]) {
        keep_current = 1; /* nothing faster to be had than what the slot holds */
        g_last.chosen_ms = t_cur;
    } else if (i_best >= 0 && i_worst >= 0 && t_of[i_worst] > 1.08f * t_of[i_best]) {
        const float t_good = t_of[i_best];
        if (chunk == bytes && !held[i_best].vmm) { /* (boxes whose spectra are chunk-sized: the chunk is the buffer) */
            chosen = held[i_best];
            held[i_best].ptr = held[i_best].vmm = NULL;
            g_last.chosen_ms = t_good;
        } else {
            /* ---- phase 2 */
            cand_free(&held[i_best]);
            used -= chunk;
            while (n_held < MAXH && used + bytes <= budget) {
                if (wall_ms_now() > deadline) {
                    timed_out = 1;
                    break;
                }
                cand p;
                /* always hipMalloc: a buffer mapped through hipMemMap runs the passes of the loop slower than
                 * one from hipMalloc even where the probe times the pair fast (profiles/r06_placement_vmm.txt) */
                if (cand_alloc(&p, bytes, bytes, 0)) break;
                used += bytes;
                if (used > peak) peak = used;
                float t = 0.f;
                const int st = c21hip_probe_pass_y2(partner, (float *)p.ptr, nx, ny, nz, reps, &t, stream);
                g_last.probes++;
                if (trace) fprintf(stderr, "[place] slot %d exact-size candidate: %.4f ms\n", slot_new, t);
                if (!st && t > 0.f && t < 1.04f * t_good) {
                    chosen = p;
                    g_last.chosen_ms = t;
                    break;
                }
                held[n_held++] = p;
                if (st) break;
            }
        }
    }
    for (int i = 0; i < n_held; i++) cand_free(&held[i]);
    if (lock_fd >= 0) {
        (void)flock(lock_fd, LOCK_UN);
        close(lock_fd);
    }
    g_last.held_gb = (double)peak / 1073741824.;
    g_walk_ms_total += wall_ms_now() - t0;
    if (keep_current) {
        float *p = plain(rec, slot_new, bytes, partner, PL_PLACED, t0);
        return p;
    }
    if (!chosen.ptr) {
        if (rec) rec->failed_bytes = bytes; /* (not repeated for this size, whether nothing differed or time ran out) */
        return plain(rec, slot_new, bytes, partner, timed_out ? PL_TIME : PL_NOTHING, t0);
    }
    if (chosen.vmm ? c21hip_ws_adopt_vmm(slot_new, chosen.vmm, bytes) : c21hip_ws_adopt(slot_new, chosen.ptr, bytes)) {
        cand_free(&chosen);
        return plain(rec, slot_new, bytes, partner, PL_NOTHING, t0);
    }
    if (rec) rec->bytes = bytes, rec->partner = partner, rec->ptr = chosen.ptr, rec->decided = 1, rec->mode_gen = g_mode_gen;
    g_last.outcome = PL_PLACED, g_last.slot = slot_new;
    g_last.wall_ms = wall_ms_now() - t0;
    return (float *)chosen.ptr;
}
