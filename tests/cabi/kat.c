/* A plain-C client of include/gorder_hip.h — what a Rust/C/C++ host does through FFI, without Python.
 *
 * Reproduces the reference's unit test of the sample arithmetic (test_calc_sch, src/analysis/mod.rs:94-105):
 * atoms at (1.7, 2.1, 9.7) and (1.9, 2.4, 0.8) in a 10-nm box, normal z  ->  S = 0.8544775, then a second
 * molecule type and leaflets by manual flags to touch more of the ABI.
 * Exit code: 0 = results as expected, 77 = no GPU (the library has no CPU fallback), 1 = wrong result. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gorder_hip.h"

int main(void) {
    /* two molecule types: A = 1 molecule with 1 bond (atoms 0-1), B = 2 molecules with 1 bond each (2-3, 4-5) */
    const uint32_t bonds_a[2] = {0, 1};
    const uint32_t bonds_b[4] = {2, 3, 4, 5};
    gorder_moltype_t types[2];
    memset(types, 0, sizeof(types));
    types[0].n_molecules = 1; types[0].n_bond_types = 1; types[0].bonds = bonds_a;
    types[1].n_molecules = 2; types[1].n_bond_types = 1; types[1].bonds = bonds_b;
    gorder_tables_t t;
    memset(&t, 0, sizeof(t));
    t.n_atoms = 6;
    t.n_molecule_types = 2;
    t.molecule_types = types;
    t.handle_pbc = 1;
    t.normal[2] = 1.0f;
    t.leaflets.method = GORDER_LEAFLETS_MANUAL;

    gorder_hip_handle *h = NULL;
    int st = gorder_hip_create(&t, &h);
    if (st == GORDER_ERR_NO_DEVICE) {
        printf("no device: %s\n", gorder_hip_strerror(st));
        gorder_hip_destroy(h);
        return 77;
    }
    if (st != GORDER_OK) { printf("create failed: %d %s\n", st, gorder_hip_last_error_message(h)); return 1; }
    if (gorder_hip_n_accumulators(h) != 2) { printf("n_acc\n"); return 1; }

    const uint8_t flags[3] = {0, 0, 1};          /* molecule 0 and 1 upper, molecule 2 lower */
    st = gorder_hip_set_manual_leaflets(h, flags, 0);
    if (st != GORDER_OK) { printf("manual leaflets: %d\n", st); return 1; }

    /* one frame; the bonds of type B are parallel (S = 1) and perpendicular (S = -0.5) to z */
    const float xyz[18] = {1.7f, 2.1f, 9.7f, 1.9f, 2.4f, 0.8f,   3.0f, 3.0f, 3.0f, 3.0f, 3.0f, 3.4f,
                           5.0f, 5.0f, 5.0f, 5.3f, 5.0f, 5.0f};
    const float box[9] = {10, 0, 0, 0, 10, 0, 0, 0, 10};
    const uint64_t frame_index[1] = {0};
    st = gorder_hip_submit_host(h, xyz, box, frame_index, 1);
    if (st != GORDER_OK) { printf("submit: %d %s\n", st, gorder_hip_last_error_message(h)); return 1; }

    int64_t sums[6];
    uint64_t counts[6], n_frames = 0;
    st = gorder_hip_finish(h, sums, counts, NULL, NULL, &n_frames);
    if (st != GORDER_OK) { printf("finish: %d %s\n", st, gorder_hip_last_error_message(h)); return 1; }
    gorder_hip_destroy(h);

    printf("S(type A) = %.7f  [reference: 0.8544775]\n", (double)sums[0] / 1e6);
    printf("type B: total %lld / %llu, upper %lld / %llu, lower %lld / %llu\n", (long long)sums[1],
           (unsigned long long)counts[1], (long long)sums[3], (unsigned long long)counts[3], (long long)sums[5],
           (unsigned long long)counts[5]);
    int ok = n_frames == 1 && counts[0] == 1 && llabs(sums[0] - 854478) <= 1;
    ok = ok && counts[1] == 2 && sums[1] == 500000;              /* 1.0 + (-0.5) in ticks */
    ok = ok && counts[2] == 1 && counts[4] == 0;                 /* type A: its molecule is upper */
    ok = ok && counts[3] == 1 && sums[3] == 1000000 && counts[5] == 1 && sums[5] == -500000;
    printf(ok ? "OK\n" : "MISMATCH\n");
    return ok ? 0 : 1;
}
