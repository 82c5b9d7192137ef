/*
 * params.c -- process-global parameter pointers of the drop-in ABI.
 *
 * reference: src/py21cmfast/src/InputParameters.c:11-90.  py21cmfast installs the
 * five cffi parameter structs once per configuration through
 * Broadcast_struct_global_all (drivers/_global_initialization.py:98-112) and keeps the
 * backing memory alive; the Compute* entry points then read them through these
 * globals.  CosmoTables is the one struct that is deep-copied.  Unlike the reference
 * (which copies the tables only the first time) every broadcast refreshes the copy
 * and invalidates the power-spectrum state, so a changed sigma_8 / cosmology is never missed.
 */
#include <stdlib.h>
#include <string.h>

#include "c21cm_abi.h"

void free_ps(void); /* cosmology.c */

SimulationOptions *simulation_options_global = NULL;
MatterOptions *matter_options_global = NULL;
CosmoParams *cosmo_params_global = NULL;
AstroParams *astro_params_global = NULL;
AstroOptions *astro_options_global = NULL;
CosmoTables *cosmo_tables_global = NULL;
ConfigSettings config_settings = {1.0, false, NULL, NULL};

static Table1D *clone_table(const Table1D *src) {
    if (!src || src->size <= 0 || !src->x_values || !src->y_values) return NULL;
    Table1D *t = (Table1D *)malloc(sizeof(Table1D));
    if (!t) return NULL;
    t->size = src->size;
    t->x_values = (double *)malloc(sizeof(double) * (size_t)src->size);
    t->y_values = (double *)malloc(sizeof(double) * (size_t)src->size);
    if (!t->x_values || !t->y_values) {
        free(t->x_values);
        free(t->y_values);
        free(t);
        return NULL;
    }
    memcpy(t->x_values, src->x_values, sizeof(double) * (size_t)src->size);
    memcpy(t->y_values, src->y_values, sizeof(double) * (size_t)src->size);
    return t;
}

static void free_table(Table1D *t) {
    if (!t) return;
    free(t->x_values);
    free(t->y_values);
    free(t);
}

/* reference: InputParameters.c:64-80 */
void Free_cosmo_tables_global(void) {
    if (!cosmo_tables_global) return;
    free_table(cosmo_tables_global->transfer_density);
    free_table(cosmo_tables_global->transfer_vcb);
    free(cosmo_tables_global);
    cosmo_tables_global = NULL;
}

/* What the power-spectrum state (sigma normalisation, transfer function, the sigma(M) spline keyed
 * on it) depends on: the cosmology, the matter options and the tables' normalisation / contents.
 * py21cmfast broadcasts the structs before EVERY Compute* call; dropping the state each time made
 * every snapshot rebuild the 416-knot sigma(M) spline.  It is dropped only when this
 * fingerprint changes (FNV-1a over the bytes; padding that differs only costs a rebuild). */
static unsigned long long fnv(unsigned long long h, const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}
static unsigned long long ps_fingerprint(const MatterOptions *mo, const CosmoParams *cp,
                                         const CosmoTables *ct) {
    unsigned long long h = 14695981039346656037ull;
    if (cp) h = fnv(h, cp, sizeof(*cp));
    if (mo) h = fnv(h, mo, sizeof(*mo));
    if (ct) {
        h = fnv(h, &ct->ps_norm, sizeof(ct->ps_norm));
        h = fnv(h, &ct->USE_SIGMA_8, sizeof(ct->USE_SIGMA_8));
        h = fnv(h, &ct->V_CB_AVG, sizeof(ct->V_CB_AVG));
        const Table1D *tabs[2] = {ct->transfer_density, ct->transfer_vcb};
        for (int i = 0; i < 2; i++)
            if (tabs[i] && tabs[i]->size > 0 && tabs[i]->x_values && tabs[i]->y_values) {
                h = fnv(h, &tabs[i]->size, sizeof(tabs[i]->size));
                h = fnv(h, tabs[i]->x_values, sizeof(double) * (size_t)tabs[i]->size);
                h = fnv(h, tabs[i]->y_values, sizeof(double) * (size_t)tabs[i]->size);
            }
    }
    return h ? h : 1;
}
/* every byte of the five installed parameter structs (callers may change them in place between calls): the key of
 * state that is valid for one configuration only (abi_compute.c: the prepared tables a sharded ComputeTsBox hands
 * from its first phase to its second) */
unsigned long long c21_params_fingerprint(void) {
    unsigned long long h = ps_fingerprint(matter_options_global, cosmo_params_global, cosmo_tables_global);
    if (simulation_options_global) h = fnv(h, simulation_options_global, sizeof(*simulation_options_global));
    if (astro_params_global) h = fnv(h, astro_params_global, sizeof(*astro_params_global));
    if (astro_options_global) h = fnv(h, astro_options_global, sizeof(*astro_options_global));
    return h;
}
static unsigned long long g_ps_print; /* 0: nothing broadcast yet */

void Broadcast_struct_global_all(SimulationOptions *simulation_options,
                                 MatterOptions *matter_options, CosmoParams *cosmo_params,
                                 AstroParams *astro_params, AstroOptions *astro_options,
                                 CosmoTables *cosmo_tables) {
    simulation_options_global = simulation_options;
    matter_options_global = matter_options;
    cosmo_params_global = cosmo_params;
    astro_params_global = astro_params;
    astro_options_global = astro_options;
    /* The power-spectrum state (sigma normalisation, EH parameters, the sigma(M) spline keyed on
     * it) belongs to the cosmology that was broadcast before: drop it, the next Compute* call
     * (or the caller's own init_ps, as py21cmfast does) rebuilds it for the new structs. */
    {
        const unsigned long long print = ps_fingerprint(matter_options, cosmo_params, cosmo_tables);
        const int same = (print == g_ps_print) && cosmo_tables_global && cosmo_tables;
        if (print != g_ps_print) free_ps();
        g_ps_print = print;
        /* unchanged: the deep copy made at the last broadcast stays (init_ps's CLASS splines point
         * into it), a changed one is replaced together with the state that was built on it */
        if (same) return;
    }
    Free_cosmo_tables_global();
    if (!cosmo_tables) return;
    cosmo_tables_global = (CosmoTables *)calloc(1, sizeof(CosmoTables));
    if (!cosmo_tables_global) return;
    cosmo_tables_global->ps_norm = cosmo_tables->ps_norm;
    cosmo_tables_global->USE_SIGMA_8 = cosmo_tables->USE_SIGMA_8;
    cosmo_tables_global->V_CB_AVG = cosmo_tables->V_CB_AVG;
    if (matter_options && matter_options->POWER_SPECTRUM == C21CM_PS_CLASS) {
        cosmo_tables_global->transfer_density = clone_table(cosmo_tables->transfer_density);
        if (matter_options->V_CB_MODEL == C21CM_VCB_FLUCTS)
            cosmo_tables_global->transfer_vcb = clone_table(cosmo_tables->transfer_vcb);
    }
}

void Broadcast_struct_global_noastro(SimulationOptions *simulation_options,
                                     MatterOptions *matter_options, CosmoParams *cosmo_params) {
    simulation_options_global = simulation_options;
    matter_options_global = matter_options;
    cosmo_params_global = cosmo_params;
    {   /* as above; the tables of the last full broadcast stay in place */
        const unsigned long long print = ps_fingerprint(matter_options, cosmo_params, cosmo_tables_global);
        if (print != g_ps_print) free_ps();
        g_ps_print = print;
    }
}
